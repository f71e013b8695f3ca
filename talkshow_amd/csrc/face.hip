// Kernels of the face generator that are not GEMM-shaped (everything GEMM-shaped runs on conv_gemm_f32):
//   w2v_conv0_*      wav2vec2 feature-extractor layer 0: Conv1d(1,512,k10,s5,no bias) + GroupNorm(512,512) + GELU
//                    (HF Wav2Vec2GroupNormConvLayer; called at nets/spg/wav2vec.py:92).  Bandwidth bound: the conv is
//                    computed in the apply pass (10 MAC/output) and never stored un-normalised; the GroupNorm statistics
//                    come from the waveform's second moments (65 sums per clip), not from a pass over the 512 channels.
//   lerp_ln          linear_interpolation 50->30 fps (nets/spg/wav2vec.py:64-70) fused with the feature-projection
//                    LayerNorm(512) (HF Wav2Vec2FeatureProjection; :107)
//   layernorm_rows   nn.LayerNorm over channels (+ post-norm residual, ReLU): encoder LNs and nets/layers.py:142-151
//   attention        fused QK^T -> online soft-max -> PV of HF eager_attention_forward, one workgroup per 64 queries of a (clip, head)
//   fill_id          id_mlp(one-hot id) broadcast over time and concatenated (nets/spg/s2g_face.py:127-130)
#include "kernels.h"

namespace ts {

constexpr int C0_TB = 128;   // output frames per block in the conv0 kernels

__device__ inline float gelu_erf(float v) { return gelu_fast(v); }   // kernels.h: libm's erf algorithm, branch-free

// partial sums of conv0 output per (clip, time block, channel): grid (tblocks, B), 256 threads x 2 channels
__global__ __launch_bounds__(256) void w2v_conv0_stats_kernel(const float *__restrict__ wav, int N, int L0,
                                                              const float *__restrict__ w, double2 *__restrict__ part,
                                                              int C) {
    __shared__ float sw[C0_TB * 5 + 16];
    const int b = blockIdx.y, tb = blockIdx.x, t0 = tb * C0_TB;
    const int nt = min(C0_TB, L0 - t0);
    for (int i = threadIdx.x; i < nt * 5 + 5; i += 256) {
        const int idx = t0 * 5 + i;
        sw[i] = idx < N ? wav[(long)b * N + idx] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float wk[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) wk[k] = w[c * 10 + k];
        double s = 0.0, s2 = 0.0;
        for (int t = 0; t < nt; ++t) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) v = fmaf(wk[k], sw[t * 5 + k], v);
            s += v;
            s2 += (double)v * v;
        }
        part[((long)b * gridDim.x + tb) * C + c] = double2{s, s2};
    }
}

// fixed-order reduction over the time blocks -> (mean, rstd) per (clip, channel)
__global__ void w2v_gn_finalize_kernel(const double2 *__restrict__ part, int ntb, int C, int L0, float2 *__restrict__ stats,
                                       int BC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C;
    double s = 0.0, s2 = 0.0;
    for (int t = 0; t < ntb; ++t) {
        const double2 v = part[((long)b * ntb + t) * C + c];
        s += v.x;
        s2 += v.y;
    }
    const double mean = s / L0;
    double var = s2 / L0 - mean * mean;
    if (var < 0) var = 0;
    stats[i] = float2{(float)mean, (float)(1.0 / sqrt(var + 1e-5))};
}

// recompute conv0, normalise, GELU, store NLC (B, L0, C)
__global__ __launch_bounds__(256) void w2v_conv0_apply_kernel(const float *__restrict__ wav, int N, int L0,
                                                              const float *__restrict__ w, const float2 *__restrict__ stats,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              float *__restrict__ out, int C) {
    __shared__ float sw[C0_TB * 5 + 16];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_TB;
    const int nt = min(C0_TB, L0 - t0);
    for (int i = threadIdx.x; i < nt * 5 + 5; i += 256) {
        const int idx = t0 * 5 + i;
        sw[i] = idx < N ? wav[(long)b * N + idx] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float wk[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) wk[k] = w[c * 10 + k];
        const float2 st = stats[b * C + c];
        const float g = gamma[c], be = beta[c];
        for (int t = 0; t < nt; ++t) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) v = fmaf(wk[k], sw[t * 5 + k], v);
            v = (v - st.x) * st.y * g + be;
            out[((long)b * L0 + t0 + t) * C + c] = gelu_erf(v);
        }
    }
}

// ---- the same statistics from the input's second moments.  conv0's output is linear in its 10-sample window: y_c[t] = sum_k w[c][k] x[5 t + k],
// so over a clip  sum_t y_c = sum_k w[c][k] S[k]  and  sum_t y_c^2 = sum_{k,k'} w[c][k] w[c][k'] R[k][k']  with S[k] = sum_t x[5 t + k] and
// R[k][k'] = sum_t x[5 t + k] x[5 t + k'] — 65 numbers per clip (10 + 55, R is symmetric) instead of 512 x 2, and one pass over the waveform that
// does 65 products per frame instead of 5 120 MACs.  Products of two floats are exact in double and all sums run in double in a fixed order: the
// statistics are those of the exact convolution (the fp32 rounding of y in the direct form moves them by ~1e-8 relative; tests bound the face
// generator against the reference either way).  Statistics pass 0.47 -> 0.07 ms per face batch of 64 (56 + 9.5 + 4.7 us for the three kernels). ----
constexpr int C0_MB = 1024;                 // frames per block of the moments kernel (20 KB of LDS: seven blocks per CU)
constexpr int C0_NQ = 65;                   // S[0..9], then R[k][k'] for k <= k' row by row
__global__ __launch_bounds__(256) void w2v_conv0_moments_kernel(const float *__restrict__ wav, int N, int L0, double *__restrict__ part) {
    __shared__ float sw[C0_MB * 5 + 16];
    __shared__ double red[3][C0_NQ];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_MB;
    const int nt = min(C0_MB, L0 - t0);
    for (int i = threadIdx.x; i < nt * 5 + 5; i += 256) {
        const int idx = t0 * 5 + i;
        sw[i] = idx < N ? wav[(long)b * N + idx] : 0.f;
    }
    __syncthreads();
    const int q = threadIdx.x % C0_NQ, slice = threadIdx.x / C0_NQ;   // 195 threads: quantity q over every third frame
    if (slice < 3) {
        int k = q, k2 = -1;                 // q < 10: S[q]
        if (q >= 10) {                      // pair number q - 10 in the order (0,0) (0,1) .. (0,9) (1,1) ..
            int r = q - 10;
            k = 0;
            while (r >= 10 - k) {
                r -= 10 - k;
                ++k;
            }
            k2 = k + r;
        }
        // eight independent partial sums: the loop is a chain of LDS round trips + one dependent double add otherwise (3 waves per SIMD)
        const float *pa = sw + k, *pb = k2 < 0 ? nullptr : sw + k2;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int t = slice;
        for (; t + 21 < nt; t += 24) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = pa[(t + 3 * u) * 5];
                vb[u] = pb ? pb[(t + 3 * u) * 5] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += (double)va[u] * (double)vb[u];
        }
        for (; t < nt; t += 3) acc[0] += (double)pa[t * 5] * (pb ? (double)pb[t * 5] : 1.0);
        red[slice][q] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    if (threadIdx.x < C0_NQ) part[((long)b * gridDim.x + blockIdx.x) * C0_NQ + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x];
}

// fixed-order sum over the blocks of a clip -> mom[b][65]
__global__ void w2v_moments_reduce_kernel(const double *__restrict__ part, int nblk, double *__restrict__ mom, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / C0_NQ, q = i - b * C0_NQ;
    double s = 0.0;
    for (int t = 0; t < nblk; ++t) s += part[((long)b * nblk + t) * C0_NQ + q];
    mom[i] = s;
}

// (mean, rstd) of channel c of clip b from the clip's moments
__global__ void w2v_gn_from_moments_kernel(const double *__restrict__ mom, const float *__restrict__ w, int C, int L0, float2 *__restrict__ stats, int BC) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C;
    const double *m = mom + (long)b * C0_NQ;
    double wk[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) wk[k] = (double)w[c * 10 + k];
    double s = 0.0, s2 = 0.0;
    int q = 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        s += wk[k] * m[k];
#pragma unroll
        for (int k2 = k; k2 < 10; ++k2, ++q) s2 += (k2 == k ? 1.0 : 2.0) * wk[k] * wk[k2] * m[q];
    }
    const double mean = s / L0;
    double var = s2 / L0 - mean * mean;
    if (var < 0) var = 0;
    stats[i] = float2{(float)mean, (float)(1.0 / sqrt(var + 1e-5))};
}

hipError_t launch_w2v_conv0(const float *wav, int B, int N, int L0, const float *w, const float *gamma, const float *beta,
                            double2 *part, float2 *stats, float *out, int C, hipStream_t s) {
    const int ntb = (L0 + C0_TB - 1) / C0_TB;
    if (knobs().w2v_moments) {   // `part` holds B x ntb x C double2: room for B x nblk x 65 + B x 65 doubles many times over
        const int nblk = (L0 + C0_MB - 1) / C0_MB;
        double *pm = reinterpret_cast<double *>(part), *mom = pm + (size_t)B * nblk * C0_NQ;
        hipLaunchKernelGGL(w2v_conv0_moments_kernel, dim3(nblk, B), dim3(256), 0, s, wav, N, L0, pm);
        hipLaunchKernelGGL(w2v_moments_reduce_kernel, dim3((B * C0_NQ + 255) / 256), dim3(256), 0, s, pm, nblk, mom, B * C0_NQ);
        hipLaunchKernelGGL(w2v_gn_from_moments_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, mom, w, C, L0, stats, B * C);
    } else {
        hipLaunchKernelGGL(w2v_conv0_stats_kernel, dim3(ntb, B), dim3(256), 0, s, wav, N, L0, w, part, C);
        hipLaunchKernelGGL(w2v_gn_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, part, ntb, C, L0, stats, B * C);
    }
    hipLaunchKernelGGL(w2v_conv0_apply_kernel, dim3(ntb, B), dim3(256), 0, s, wav, N, L0, w, stats, gamma, beta, out, C);
    return hipGetLastError();
}

// ---- row-wise LayerNorm: one wavefront per row, C = 64 * CPL -----------------------------------------------------
template <int CPL>
__device__ inline void ln_row(float (&v)[CPL], const float *gamma, const float *beta, int lane, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) s += v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.0f / (64 * CPL));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q * (1.0f / (64 * CPL)) + eps);
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = (v[i] - mean) * rstd * gamma[lane + 64 * i] + beta[lane + 64 * i];
}

template <int CPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float *__restrict__ x, int ldx, long M,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ post_res, int ldr, int relu,
                                                             float *__restrict__ out, int ldo) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float v[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = x[m * ldx + lane + 64 * i];
    ln_row<CPL>(v, gamma, beta, lane, 1e-5f);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        float y = v[i];
        if (post_res) y += post_res[m * ldr + lane + 64 * i];
        if (relu) y = y > 0.f ? y : 0.f;
        out[m * ldo + lane + 64 * i] = y;
    }
}

hipError_t launch_layernorm_rows(const float *x, int ldx, long M, int C, const float *gamma, const float *beta,
                                 const float *post_res, int ldr, int relu, float *out, int ldo, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    switch (C) {
        case 64: hipLaunchKernelGGL(layernorm_rows_kernel<1>, grid, block, 0, s, x, ldx, M, gamma, beta, post_res, ldr, relu, out, ldo); break;
        case 256: hipLaunchKernelGGL(layernorm_rows_kernel<4>, grid, block, 0, s, x, ldx, M, gamma, beta, post_res, ldr, relu, out, ldo); break;
        case 512: hipLaunchKernelGGL(layernorm_rows_kernel<8>, grid, block, 0, s, x, ldx, M, gamma, beta, post_res, ldr, relu, out, ldo); break;
        case 768: hipLaunchKernelGGL(layernorm_rows_kernel<12>, grid, block, 0, s, x, ldx, M, gamma, beta, post_res, ldr, relu, out, ldo); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- time interpolation (align_corners=False) + LayerNorm(512) --------------------------------------------------
__global__ __launch_bounds__(256) void lerp_ln_kernel(const float *__restrict__ x, int Lin, int T, long M,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      float *__restrict__ out) {
    constexpr int CPL = 8, C = 512;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(m / T), j = (int)(m - (long)b * T);
    const float scale = (float)Lin / (float)T;
    float src = scale * ((float)j + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)floorf(src);
    const int i1 = min(i0 + 1, Lin - 1);
    const float l1 = src - (float)i0, l0 = 1.0f - l1;
    const float *r0 = x + ((long)b * Lin + i0) * C, *r1 = x + ((long)b * Lin + i1) * C;
    float v[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = r0[lane + 64 * i] * l0 + r1[lane + 64 * i] * l1;
    ln_row<CPL>(v, gamma, beta, lane, 1e-5f);
#pragma unroll
    for (int i = 0; i < CPL; ++i) out[m * C + lane + 64 * i] = v[i];
}
hipError_t launch_lerp_ln(const float *x, int B, int Lin, int T, const float *gamma, const float *beta, float *out,
                          hipStream_t s) {
    const long M = (long)B * T;
    hipLaunchKernelGGL(lerp_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, Lin, T, M, gamma, beta, out);
    return hipGetLastError();
}

// ---- fused attention of one wav2vec2 encoder layer (HF eager_attention_forward: softmax(Q K^T * d^-0.5) V, wav2vec.py:76-143) ----
// One workgroup = 64 queries of one (clip, head); wave w owns queries 16 w .. 16 w + 15 and walks the keys in tiles of 64 that all
// four waves share through LDS (K and V as [key][d], rows pitched 68 floats).  Both products run on v_mfma_f32_16x16x4_f32 with the KEYS as the rows of the first product:
//     S^T[key][query] = K Q^T        lane (li, lg) ends up with keys 4 lg .. 4 lg + 3 of each 16-key block for query li
//     O^T[d][query]  += V^T P^T      ... which is exactly the B-operand fragment of the second product (k index = key 4 lg + e)
// so the probabilities never leave their registers, a query's running max / sum are two xor-shuffles across the four lane groups,
// and the (B, heads, T, T) score tensor of the launch-per-op form (QK^T GEMM -> softmax kernel -> V transpose kernel -> PV GEMM:
// 0.77 GB written and read back per layer at batch 64) does not exist.  Online soft-max over the key tiles (running max m, sum l,
// O rescaled by exp(m_old - m_new)): any T, no 2^31-entry score buffer.  fp32 throughout; the scale 2^-3 is exact.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IC4 { static constexpr int value = V; };
constexpr int ATT_P = 68;
// all-reduce across the four 16-lane rows of a wave with the gfx950 row-swap instructions (VALU; a __shfl_xor is a ds_bpermute: an LDS
// round trip on the soft-max's critical path): v_permlane16_swap(x, x) -> {rows (0,0,2,2), rows (1,1,3,3)}, v_permlane32_swap(x, x) ->
// {lower half twice, upper half twice}
template <class Op> __device__ __forceinline__ float rows_allreduce(float x, Op op) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__global__ __launch_bounds__(256, 2) void attention_kernel(const float *__restrict__ qkv, int T, int HID, int heads, int nz, float scale,
                                                           float *__restrict__ out) {
    __shared__ float Ks[64 * ATT_P];
    __shared__ float Vs[64 * ATT_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    // workgroup id -> (clip-head z, query tile): consecutive ids go to consecutive XCDs, so the nq query tiles of one (clip, head)
    // take ids that are congruent mod 8 — they run on ONE XCD, back to back, and find K / V of their (clip, head) in its L2
    // (with the query tile as the fastest grid index every tile of a (clip, head) pulled its own copy over the fabric:
    // 708 MB per launch at batch 64 against 236 MB of compulsory traffic)
    const int nq = (T + 63) >> 6;
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
    const int z = (grp / nq) * 8 + xcd;
    if (z >= nz) return;
    const int b = z / heads, h = z - b * heads;
    const int q0 = (grp % nq) * 64 + wave * 16;
    const long ld = 3L * HID;
    const float *base = qkv + (long)b * T * ld + h * 64;
    // soft-max in base 2: exp(s * scale - max) = 2^(s * scale * log2 e - max'), one v_exp_f32 per probability instead of expf's
    // range reduction (32 of them per lane and key tile: as many VALU slots as the tile's MFMAs have issue slots)
    const float scale2 = scale * 1.44269504088896341f;
    // Q fragments (pre-multiplied by scale * log2 e), B operand of the first product: lane (li, lg) holds Q[q0 + li][16 qs + 4 lg + e]; rows past T are clamped (computed, never stored)
    f32x4 qf[4];
    {
        const int qrow = q0 + li < T ? q0 + li : T - 1;
#pragma unroll
        for (int qs = 0; qs < 4; ++qs) qf[qs] = *reinterpret_cast<const f32x4 *>(base + (long)qrow * ld + 16 * qs + 4 * lg) * scale2;
    }
    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const bool live = q0 < T;   // wave-uniform: this wave has at least one real query (it still stages K / V and meets the barriers)
    for (int k0 = 0; k0 < T; k0 += 64) {
        __syncthreads();   // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i, col = (tid & 15) * 4, key = k0 + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                kv = *reinterpret_cast<const f32x4 *>(base + (long)key * ld + HID + col);
                vv = *reinterpret_cast<const f32x4 *>(base + (long)key * ld + 2 * HID + col);
            }
            *reinterpret_cast<f32x4 *>(&Ks[row * ATT_P + col]) = kv;
            *reinterpret_cast<f32x4 *>(&Vs[row * ATT_P + col]) = vv;
        }
        __syncthreads();
        // The products of one key tile for NKB real 16-key blocks (compile-time: the MFMA stream has no branches in it).  The key block /
        // d block is the INNER loop of both products: four independent accumulators take turns, so an MFMA never waits for its
        // predecessor's result (16 in a row on one accumulator issue every 40 cycles, not 32).  Key blocks wholly beyond T (the last
        // tile of a 300-frame clip has three real blocks) are not multiplied, nor are waves whose 16 queries all lie beyond T.
        auto tile = [&](auto NKBc, auto RAGc) {
            constexpr int NKB = decltype(NKBc)::value;
            constexpr bool RAGGED = decltype(RAGc)::value != 0;   // the tile reaches beyond T: its padding keys are masked
            f32x4 sacc[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qs = 0; qs < 4; ++qs) {
                f32x4 kf[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) kf[kb] = *reinterpret_cast<const f32x4 *>(&Ks[(kb * 16 + li) * ATT_P + 16 * qs + 4 * lg]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb][e], qf[qs][e], sacc[kb], 0, 0, 0);
            }
            // scale, mask the padding keys, online soft-max of query li (this lane's keys: k0 + 16 kb + 4 lg + r)
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RAGGED && (k0 + kb * 16 + 4 * lg + r) >= T) sacc[kb][r] = -INFINITY;
                    mx = fmaxf(mx, sacc[kb][r]);
                }
            mx = rows_allreduce(mx, [](float a, float b) { return fmaxf(a, b); });
            const float m_new = fmaxf(m, mx);          // finite: every tile holds at least one real key
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);       // first tile: 2^-inf = 0
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] - m_new);
                    sacc[kb][r] = pv;
                    rs += pv;
                }
            rs = rows_allreduce(rs, [](float a, float b) { return a + b; });
            l = l * alpha + rs;
            m = m_new;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] *= alpha;
            // O^T += V^T P^T; the B operand is the probability registers as they are.  The A operand V^T[d = li][key = 4 lg + e] is read
            // from V as it was staged ([key][d], 16 consecutive d per lane group: four ds_read_b32, rows 4 lg + e of a 68-float pitch
            // land 16 banks apart for lg and lg + 1: conflict-free) — a transposed copy of V would need 16 scattered ds_write_b32 per
            // thread and tile, 8 lanes to a bank (measured: 63 % of the LDS cycles were conflicts)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float va[4];
#pragma unroll
                    for (int db = 0; db < 4; ++db) va[db] = Vs[(kb * 16 + 4 * lg + e) * ATT_P + db * 16 + li];
#pragma unroll
                    for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[db], sacc[kb][e], o[db], 0, 0, 0);
                }
        };
        if (live) {
            const int nkb = (T - k0 + 15) >> 4;
            if (k0 + 64 <= T) tile(IC4<4>{}, IC4<0>{});
            else if (nkb >= 4) tile(IC4<4>{}, IC4<1>{});
            else if (nkb == 3) tile(IC4<3>{}, IC4<1>{});
            else if (nkb == 2) tile(IC4<2>{}, IC4<1>{});
            else tile(IC4<1>{}, IC4<1>{});
        }
    }
    if (q0 + li < T) {
        const float inv = 1.0f / l;
        float *dst = out + ((long)b * T + q0 + li) * HID + h * 64 + 4 * lg;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4 *>(dst + db * 16) = o[db] * inv;
    }
}
// qkv (B, T, 3 HID) rows [q | k | v], heads of 64 channels -> out (B, T, HID) = concatenated heads' softmax(q k^T * scale) v
hipError_t launch_attention(const float *qkv, int B, int T, int HID, int heads, float scale, float *out, hipStream_t s) {
    if (HID != heads * 64 || B < 1 || T < 1 || (long)B * heads * ((T + 63) / 64) > (1l << 30)) return hipErrorInvalidValue;
    const int nq = (T + 63) / 64, nz = B * heads;
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)(((nz + 7) / 8) * 8 * nq)), dim3(256), 0, s, qkv, T, HID, heads, nz, scale, out);
    return hipGetLastError();
}

// ---- id channels: x[b][t][col0 + j] = bias[j] + sum_c W[j][c] * id[b][c] ----------------------------------------
__global__ void fill_id_kernel(const float *__restrict__ id, int nc, const float *__restrict__ w, const float *__restrict__ bias,
                               int nj, float *__restrict__ x, int ld, int col0, long rows, int T) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nj) return;
    const long m = i / nj;
    const int j = (int)(i - m * nj);
    const int b = (int)(m / T);
    float v = bias[j];
    for (int c = 0; c < nc; ++c) v = fmaf(w[j * nc + c], id[b * nc + c], v);
    x[m * ld + col0 + j] = v;
}
hipError_t launch_fill_id(const float *id, int nc, const float *w, const float *bias, int nj, float *x, int ld, int col0,
                          int B, int T, hipStream_t s) {
    const long n = (long)B * T * nj;
    hipLaunchKernelGGL(fill_id_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, id, nc, w, bias, nj, x, ld, col0,
                       (long)B * T, T);
    return hipGetLastError();
}

}  // namespace ts
