// VQ codebook search, gathers, the per-position sampler and small glue kernels.
//
//   vq_argmin   VectorQuantizerEMA.get_code_indices   nets/spg/vqvae_modules.py:311-319
//   gather_rows VectorQuantizerEMA.quantize + the (B,W,64)->(B,64,W) permute of VQVAE.decode (a no-op in NLC)
//               nets/spg/vqvae_modules.py:321-323, nets/spg/vqvae_1d.py:201-208
//   sample      softmax + multinomial(1) of GatedPixelCNN.generate, or the greedy argmax harness
//               nets/spg/gated_pixelcnn_v2.py:173-176
#include "kernels.h"
#include "skinny_desc.h"
#include <cstring>
#include "../../include/talkshow_hip.h"

namespace ts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// vq_argmin: one workgroup = ROWS query rows x all codes.  Queries sit in LDS; each thread walks codes
// j = tid, tid+256, ... (ascending, so a strict '<' keeps the lowest index on ties), reading the code row as
// 16-byte loads straight from L2 (the 2048x64 fp32 codebook is 512 KiB and stays cache resident).  Distance is
// evaluated in the reference's association: (|x|^2 + |e_j|^2) - 2*(x.e_j).  Block argmin = wavefront shuffle
// reduction on (distance, index) pairs + one LDS hop across the 4 waves.
// ---------------------------------------------------------------------------------------------------------------
constexpr int VQ_ROWS = 8;

__global__ __launch_bounds__(256) void vq_argmin_kernel(const float *__restrict__ x, int ldx, int M,
                                                        const float *__restrict__ cb, const float *__restrict__ csq,
                                                        int ncode, int dim, int64_t *idx, long idx_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm;                              // [VQ_ROWS][dim]
    float *xsq = sm + VQ_ROWS * dim;             // [VQ_ROWS]
    float *rd = xsq + VQ_ROWS;                   // [4][VQ_ROWS]
    int *ri = reinterpret_cast<int *>(rd + 4 * VQ_ROWS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * VQ_ROWS;
    for (int i = tid; i < VQ_ROWS * dim; i += 256) {
        int r = i / dim, c = i - r * dim;
        xs[i] = (m0 + r < M) ? x[(long)(m0 + r) * ldx + c] : 0.f;
    }
    __syncthreads();
    if (tid < VQ_ROWS) {
        float s = 0.f;
        for (int c = 0; c < dim; ++c) s += xs[tid * dim + c] * xs[tid * dim + c];
        xsq[tid] = s;
    }
    __syncthreads();

    float best[VQ_ROWS];
    int bidx[VQ_ROWS];
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) { best[r] = INFINITY; bidx[r] = 0x7fffffff; }

    for (int j = tid; j < ncode; j += 256) {
        float dot[VQ_ROWS];
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) dot[r] = 0.f;
        const float4 *e = reinterpret_cast<const float4 *>(cb + (long)j * dim);
        for (int c4 = 0; c4 < dim / 4; ++c4) {
            const float4 ev = e[c4];
#pragma unroll
            for (int r = 0; r < VQ_ROWS; ++r) {
                const float4 xv = *reinterpret_cast<const float4 *>(&xs[r * dim + c4 * 4]);
                dot[r] = fmaf(xv.x, ev.x, dot[r]);
                dot[r] = fmaf(xv.y, ev.y, dot[r]);
                dot[r] = fmaf(xv.z, ev.z, dot[r]);
                dot[r] = fmaf(xv.w, ev.w, dot[r]);
            }
        }
        const float ee = csq[j];
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) {
            const float d = (xsq[r] + ee) - 2.0f * dot[r];
            if (d < best[r]) { best[r] = d; bidx[r] = j; }
        }
    }
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) {
        float d = best[r];
        int j = bidx[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(d, off);
            const int oj = __shfl_xor(j, off);
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        if (lane == 0) { rd[wave * VQ_ROWS + r] = d; ri[wave * VQ_ROWS + r] = j; }
    }
    __syncthreads();
    if (tid < VQ_ROWS && m0 + tid < M) {
        float d = rd[tid];
        int j = ri[tid];
        for (int w = 1; w < 4; ++w) {
            const float od = rd[w * VQ_ROWS + tid];
            const int oj = ri[w * VQ_ROWS + tid];
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        idx[(long)(m0 + tid) * idx_stride] = j;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// vq_argmin, LDS-staged form (dim = 64: the VQ-VAE's embedding width).  The kernel above lets every thread pull its own code
// rows from L2 — a wave load touches 64 rows x 16 B, and a workgroup of 8 query rows reads the whole 512 KB codebook.  Here:
//   * a workgroup serves 4 x RW query rows and walks the codebook in TILES of 64 codes, staged through LDS by coalesced 16-byte
//     loads (a tile is 16 KB contiguous) into rows pitched 68 floats — lane l then reads code l of the tile with ds_read_b128,
//     conflict-free — and double-buffered: tile t + 1 is on its way while tile t is multiplied;
//   * lane = code, wave = RW query rows: the query values are WAVE-UNIFORM, so they come through the scalar unit (s_load) and
//     enter the FMAs as scalar operands — no LDS traffic, no broadcast, 64 v_fmac per (row, tile) against the lane's 64 code
//     registers;
//   * a lane keeps its running (distance, index) per row over the tiles it sees (codes l, l + 64, ...: ascending, strict '<'
//     keeps the lowest index), one wavefront reduction per row at the end (ties -> lowest index).
// Same arithmetic as above, operation for operation — dot as the c-ascending fmaf chain, (|x|^2 + |e|^2) - 2 dot, |x|^2 summed
// in c order — so the two kernels return the same index for every row (tests/test_gpu_parity.py::test_vq_argmin_lds_form).
// ---------------------------------------------------------------------------------------------------------------
constexpr int VQ_TILE = 64, VQ_DIM = 64, VQ_PITCH = VQ_DIM + 4;

template <int RW>
__global__ __launch_bounds__(256) void vq_argmin_lds_kernel(const float *__restrict__ x, int ldx, int M, const float *__restrict__ cb,
                                                            const float *__restrict__ csq, int ncode, int64_t *__restrict__ idx, long idx_stride) {
    __shared__ __attribute__((aligned(16))) float tile[2][VQ_TILE][VQ_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * (4 * RW) + wave * RW;   // this wave's first query row (wave-uniform)
    const int ntile = (ncode + VQ_TILE - 1) / VQ_TILE;

    // |x|^2 of the wave's rows, in the order the kernel above sums it (wave-uniform values: every lane computes the same number)
    float xsq[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float *xr = x + (long)(m0 + r < M ? m0 + r : 0) * ldx;
        float sq = 0.f;
        for (int c = 0; c < VQ_DIM; ++c) sq += xr[c] * xr[c];
        xsq[r] = sq;
    }
    float best[RW];
    int bidx[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { best[r] = INFINITY; bidx[r] = 0x7fffffff; }

    // staging: a tile is 64 x 64 floats = 1024 16-byte chunks, 4 per thread; chunk q -> row q / 16, columns 4 (q % 16) ..
    f32x4 st[4];
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 4;
            st[i] = t * VQ_TILE + row < ncode ? *reinterpret_cast<const f32x4 *>(cb + ((long)t * VQ_TILE + row) * VQ_DIM + (q & 15) * 4)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            *reinterpret_cast<f32x4 *>(&tile[buf][q >> 4][(q & 15) * 4]) = st[i];
        }
    };
    fetch(0);
    park(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) fetch(t + 1);   // in flight under this tile's FMAs
        f32x4 e[VQ_DIM / 4];
#pragma unroll
        for (int c4 = 0; c4 < VQ_DIM / 4; ++c4) e[c4] = *reinterpret_cast<const f32x4 *>(&tile[buf][lane][c4 * 4]);
        const int j = t * VQ_TILE + lane;
        const float ee = j < ncode ? csq[j] : 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float *xr = x + (long)(m0 + r < M ? m0 + r : 0) * ldx;   // wave-uniform address: scalar loads
            float dot = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < VQ_DIM / 4; ++c4) {
                dot = fmaf(xr[c4 * 4 + 0], e[c4][0], dot);
                dot = fmaf(xr[c4 * 4 + 1], e[c4][1], dot);
                dot = fmaf(xr[c4 * 4 + 2], e[c4][2], dot);
                dot = fmaf(xr[c4 * 4 + 3], e[c4][3], dot);
            }
            const float d = (xsq[r] + ee) - 2.0f * dot;
            if (j < ncode && d < best[r]) { best[r] = d; bidx[r] = j; }
        }
        if (t + 1 < ntile) park(buf ^ 1);   // the other buffer was last read two iterations ago, behind the barrier below
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float d = best[r];
        int j = bidx[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(d, off);
            const int oj = __shfl_xor(j, off);
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        if (lane == 0 && m0 + r < M) idx[(long)(m0 + r) * idx_stride] = j;
    }
}

hipError_t launch_vq_argmin(const float *x, int ldx, int M, const float *codebook, const float *code_sq, int ncode,
                            int dim, int64_t *idx, long idx_stride, hipStream_t stream) {
    if (dim % 4 != 0) return hipErrorInvalidValue;
    if (dim == VQ_DIM && knobs().vq_lds && (reinterpret_cast<uintptr_t>(codebook) & 15) == 0) {
        // 32 rows per workgroup once that still fills the chip, else 8 (the batch-of-32 call: 2 400 rows)
        if (M >= 32 * 512) hipLaunchKernelGGL(vq_argmin_lds_kernel<8>, dim3((M + 31) / 32), dim3(256), 0, stream, x, ldx, M, codebook, code_sq, ncode, idx, idx_stride);
        else hipLaunchKernelGGL(vq_argmin_lds_kernel<2>, dim3((M + 7) / 8), dim3(256), 0, stream, x, ldx, M, codebook, code_sq, ncode, idx, idx_stride);
        return hipGetLastError();
    }
    size_t smem = (VQ_ROWS * dim + VQ_ROWS + 4 * VQ_ROWS) * sizeof(float) + 4 * VQ_ROWS * sizeof(int);
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((M + VQ_ROWS - 1) / VQ_ROWS), dim3(256), smem, stream, x, ldx, M,
                       codebook, code_sq, ncode, dim, idx, idx_stride);
    return hipGetLastError();
}

__global__ void row_sqnorm_kernel(const float *e, int n, int dim, float *out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int c = 0; c < dim; ++c) s += e[(long)j * dim + c] * e[(long)j * dim + c];
    out[j] = s;
}
hipError_t launch_row_sqnorm(const float *e, int n, int dim, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, e, n, dim, out);
    return hipGetLastError();
}

// one wave per row, 16-byte lanes
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ table, int ld_table,
                                                          const int64_t *__restrict__ idx, long idx_stride, int M,
                                                          int width, float *__restrict__ out, int ldo, int nrows) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = idx[(long)m * idx_stride];
    float4 *dst = reinterpret_cast<float4 *>(out + (long)m * ldo);
    if (r < 0 || r >= nrows) {   // nn.Embedding would raise IndexError; here: memory-safe and loud (a row of NaNs)
        const float q = __builtin_nanf("");
        for (int c = lane; c < width / 4; c += 64) dst[c] = make_float4(q, q, q, q);
        return;
    }
    const float4 *src = reinterpret_cast<const float4 *>(table + r * ld_table);
    for (int c = lane; c < width / 4; c += 64) dst[c] = src[c];
}
hipError_t launch_gather_rows(const float *table, int ld_table, int nrows, const int64_t *idx, long idx_stride, int M,
                              int width, float *out, int ldo, hipStream_t stream) {
    if (width % 4 || ld_table % 4 || ldo % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, table, ld_table, idx, idx_stride,
                       M, width, out, ldo, nrows);
    return hipGetLastError();
}

__global__ void pad_rows_kernel(const float *__restrict__ src, int lds_, int c, float *__restrict__ dst, int ldd,
                                int cpad, long M) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cpad) return;
    const long m = i / cpad;
    const int k = (int)(i - m * cpad);
    dst[m * ldd + k] = k < c ? src[m * lds_ + k] : 0.f;
}
hipError_t launch_pad_rows(const float *src, int lds_, int c, float *dst, int ldd, int cpad, long M, hipStream_t stream) {
    const long n = M * cpad;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, lds_, c, dst,
                       ldd, cpad, M);
    return hipGetLastError();
}

// Whole-body output assembly (scripts/demo.py:207-229 + data_utils/lower_body.py:68-87): per frame
//   p232 = [jaw = face[0:3] | body/hand 129 (last frame repeated / trimmed to the face length) | expression = face[3:103]]
//   out265 = [p[0:3] lp[0:15] p[3:6] lp[15:21] p[6:9] lp[21:27] p[9:12] lp[27:33] p[12:232]]
// One thread per output element; pure copies, HBM-bound (1.9 KB per frame).
struct AssembleParams {
    const float *body;   // (B, Tb, 129)
    const float *face;   // (B, Tf, 103)
    float *out;          // (B, Tf, 265)
    int B, Tb, Tf;
    float lp[33];        // the fixed lower-body pose block
};
__global__ void assemble_full_kernel(const AssembleParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)p.B * p.Tf * 265;
    if (i >= n) return;
    const int c = (int)(i % 265);
    const long bt = i / 265;
    const int t = (int)(bt % p.Tf), b = (int)(bt / p.Tf);
    // column of the 232-vector this output column copies, or -(k+1) for lower-pose constant k
    int src;
    if (c < 3) src = c;
    else if (c < 18) src = -(c - 3 + 1);
    else if (c < 21) src = c - 15;
    else if (c < 27) src = -(15 + c - 21 + 1);
    else if (c < 30) src = c - 21;
    else if (c < 36) src = -(21 + c - 30 + 1);
    else if (c < 39) src = c - 27;
    else if (c < 45) src = -(27 + c - 39 + 1);
    else src = c - 33;
    float v;
    if (src < 0) {
        v = p.lp[-src - 1];
    } else if (src < 3) {
        v = p.face[((long)b * p.Tf + t) * 103 + src];
    } else if (src < 132) {
        const int tb = t < p.Tb ? t : p.Tb - 1;
        v = p.body[((long)b * p.Tb + tb) * 129 + (src - 3)];
    } else {
        v = p.face[((long)b * p.Tf + t) * 103 + 3 + (src - 132)];
    }
    p.out[i] = v;
}
hipError_t launch_assemble_full(const float *body, const float *face, const float *lower_pose33, int B, int Tb, int Tf,
                                float *out, hipStream_t stream) {
    AssembleParams p;
    p.body = body; p.face = face; p.out = out; p.B = B; p.Tb = Tb; p.Tf = Tf;
    for (int k = 0; k < 33; ++k) p.lp[k] = lower_pose33[k];
    const long n = (long)B * Tf * 265;
    hipLaunchKernelGGL(assemble_full_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

__global__ void i64_to_i32_kernel(const int64_t *src, int *dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int)src[i];
}
hipError_t launch_i64_to_i32(const int64_t *src, int *dst, long n, hipStream_t stream) {
    hipLaunchKernelGGL(i64_to_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

// three 64-bit words (the sampler's {seed, first clip index, Philox position base}) written by a kernel whose ARGUMENTS carry
// them: arguments are copied when the launch is queued, so — unlike an asynchronous copy out of host memory — nothing on the
// host has to outlive the call, however many calls are queued behind each other on the stream
__global__ void set_words3_kernel(uint64_t *dst, uint64_t a, uint64_t b, uint64_t c) {
    dst[0] = a;
    dst[1] = b;
    dst[2] = c;
}
hipError_t launch_set_words3(uint64_t *dst, uint64_t a, uint64_t b, uint64_t c, hipStream_t stream) {
    hipLaunchKernelGGL(set_words3_kernel, dim3(1), dim3(1), 0, stream, dst, a, b, c);
    return hipGetLastError();
}

// test aid: out[i] = gate_act(v[i], p[i]) — the chain kernels' gate on given operands (tests measure it against tanh * sigmoid in float64)
__global__ void gate_act_kernel(const float *v, const float *p, float *out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gate_act(v[i], p[i]);
}
hipError_t launch_gate_act(const float *v, const float *p, float *out, long n, hipStream_t stream) {
    hipLaunchKernelGGL(gate_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, v, p, out, n);
    return hipGetLastError();
}

// measurement aid: one wave that sleeps and, every `window_ticks` of the 100 MHz wall clock, records (wall ticks, shader cycles)
// since the previous record — the shader clock the chip actually runs at while other streams load it
__global__ void clock_sample_kernel(unsigned long long *out, int n, unsigned long long window_ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long w0 = wall_clock64(), c0 = clock64();
    const unsigned long long wstart = w0;
    for (int i = 0; i < n; ++i) {
        unsigned long long w1;
        do {
            __builtin_amdgcn_s_sleep(64);
            w1 = wall_clock64();
        } while (w1 - w0 < window_ticks);
        const unsigned long long c1 = clock64();
        out[3 * i] = w1 - wstart;
        out[3 * i + 1] = w1 - w0;
        out[3 * i + 2] = c1 - c0;
        w0 = w1;
        c0 = c1;
    }
}
hipError_t launch_clock_sample(unsigned long long *out, int n, unsigned long long window_ticks, hipStream_t stream) {
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, stream, out, n, window_ticks);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// sampler: one workgroup per clip over V logits.
//   greedy   : argmax, ties -> lowest index (torch.argmax on CPU returns the first maximum).
//   sampling : inverse CDF of softmax(logits).  p_v ∝ exp(l_v - max); thread t owns the contiguous chunk
//              [t*V/256, (t+1)*V/256), sums it left to right; thread 0 prefix-sums the 256 chunk sums left to right;
//              the draw is the first index whose running sum exceeds u * total.  oracle/talkshow_oracle.py
//              (`sample_inverse_cdf`) restates exactly this summation structure AND the exponential (`det_expf` below: fp32
//              multiplies / adds only), so the draw of a given uniform is the same index bit for bit.
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t &o0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0;
}

__global__ __launch_bounds__(256) void sample_kernel(const SampleParams p) {
    __shared__ float sf[256 + 1];
    __shared__ int si[256];
    __shared__ float s_thr;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *lg = p.logits + (long)b * (p.logit_stride ? p.logit_stride : (long)p.V);

    // Every thread owns `chunk` consecutive logits [v0, v1).  For the production vocabulary (V = 2048: chunk = 8) they are
    // fetched up front with two 16-byte loads — this kernel sits on the dependent chain, and a load-per-iteration loop
    // costs one memory round trip per element.
    const int chunk = (p.V + 255) / 256;
    const int v0 = tid * chunk, v1 = min(v0 + chunk, p.V);
    const bool fast = chunk == 8 && (p.V & 7) == 0;
    float x[8];
    if (fast) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(lg + v0), hi = *reinterpret_cast<const f32x4 *>(lg + v0 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = lo[k]; x[4 + k] = hi[k]; }
    }

    if (p.logits_copy) {
        float *dst = p.logits_copy + (long)b * p.copy_stride;
        if (fast) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[v0 + k] = x[k];
        } else {
            for (int v = v0; v < v1; ++v) dst[v] = lg[v];
        }
    }

    if (p.mode == TS_TEACHER_FORCED) {
        if (tid == 0) p.tok32[(long)b * p.tok_stride] = (int)p.codes[(long)b * p.code_stride];
        return;
    }

    // ---- max / argmax (needed by both modes); ties -> lowest index ----
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (fast) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (x[k] > best) { best = x[k]; bi = v0 + k; }
    } else {
        for (int v = v0; v < v1; ++v) {
            const float t = lg[v];
            if (t > best) { best = t; bi = v; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sf[wave] = best; si[wave] = bi; }
    __syncthreads();
    best = sf[0]; bi = si[0];
    for (int w = 1; w < 4; ++w)
        if (sf[w] > best || (sf[w] == best && si[w] < bi)) { best = sf[w]; bi = si[w]; }
    __syncthreads();

    int choice = bi;
    if (p.mode != TS_SAMPLE_GREEDY) {
        float u;
        if (p.mode == TS_SAMPLE_UNIFORMS) {
            u = p.uniforms[(long)b * p.u_stride];
        } else {
            const uint64_t seed = p.dyn ? p.dyn[0] : p.seed;
            const uint64_t clip = (uint64_t)((p.dyn ? (int64_t)p.dyn[1] : p.clip_index0) + b);
            uint32_t r;
            philox4x32_10(p.position + (p.dyn ? (uint32_t)p.dyn[2] : 0u), (uint32_t)clip, (uint32_t)(clip >> 32), 0u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), r);
            u = (float)(r >> 8) * (1.0f / 16777216.0f);
        }
        float s = 0.f;
        if (fast) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s += det_expf(x[k] - best);
        } else {
            for (int v = v0; v < v1; ++v) s += det_expf(lg[v] - best);
        }
        sf[tid + 1] = s;
        __syncthreads();
        if (tid == 0) {
            float c = 0.f;
            sf[0] = 0.f;
            for (int t = 1; t <= 256; ++t) { c += sf[t]; sf[t] = c; }   // sf[t] = sum of chunks < t
            s_thr = u * c;
        }
        __syncthreads();
        const float thr = s_thr;
        // owner: the first chunk whose inclusive prefix exceeds thr (the last non-empty chunk if none does)
        const bool mine = (sf[tid] <= thr) && (thr < sf[tid + 1] || tid == 255);
        if (mine && v0 < p.V) {
            float c = sf[tid];
            int k = v1 - 1;
            if (fast) {
                bool found = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) {   // same running sum as the loop below; the first crossing is latched
                    c += det_expf(x[j] - best);
                    if (!found && c > thr) { k = v0 + j; found = true; }
                }
            } else {
                for (int v = v0; v < v1; ++v) {
                    c += det_expf(lg[v] - best);
                    if (c > thr) { k = v; break; }
                }
            }
            si[0] = k;
        } else if (mine) {
            si[0] = p.V - 1;
        }
        __syncthreads();
        choice = si[0];
    }
    if (tid == 0) {
        p.tok32[(long)b * p.tok_stride] = choice;
        p.codes[(long)b * p.code_stride] = choice;
    }
}

hipError_t launch_sample(const SampleParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(sample_kernel, dim3(p.B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ts
