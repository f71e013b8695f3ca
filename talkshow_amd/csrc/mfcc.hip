// Device audio front-end kernels (the GEMM-shaped parts — the mel projection, the DCT — run on conv_gemm_f32):
//   resample_polyphase   torchaudio.transforms.Resample (sinc_interp_hann) as a polyphase FIR
//   stft_power           reflect padding (center=True) + framing (n_fft 2048, hop) + periodic Hann window + real FFT + |X|^2, one frame
//                        per workgroup, zero-padded to a multiple of 32 bins
//   db_topdb             10*log10(clamp(x, 1e-10)) and the per-clip clamp at (max - top_db)
// Reference call site: data_utils/utils.py:148-231 (get_mfcc_ta) -> torchaudio.transforms.{Resample, MFCC}.
#include "kernels.h"

namespace ts {

// out[b][j] = sum_k kern[j % nnew][k] * xpad[(j / nnew) * norig + k],  xpad = x shifted by `width` zeros on the left
__global__ void resample_polyphase_kernel(const float *__restrict__ x, int N, const float *__restrict__ kern, int norig,
                                          int nnew, int width, int kw, float *__restrict__ out, int Nout) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Nout) return;
    const int wdw = j / nnew, ph = j - wdw * nnew;
    const float *kr = kern + ph * kw;
    const float *xb = x + (long)b * N;
    const int base = wdw * norig - width;
    float acc = 0.f;
    for (int k = 0; k < kw; ++k) {
        const int i = base + k;
        if (i >= 0 && i < N) acc = fmaf(kr[k], xb[i], acc);
    }
    out[(long)b * Nout + j] = acc;
}
// The same sum with the polyphase table and the block's input window staged in LDS (the plain kernel fetches 2 kw values per output
// through L1: 58 % of the front-end once the STFT became an FFT).  Same terms in the same order — taps that fall outside the clip
// multiply a staged zero instead of being skipped, which leaves the fma chain's value unchanged — so the samples are bit-identical.
__global__ __launch_bounds__(256) void resample_polyphase_lds_kernel(const float *__restrict__ x, int N, const float *__restrict__ kern,
                                                                     int norig, int nnew, int width, int kw, float *__restrict__ out,
                                                                     int Nout, int nwin) {
    extern __shared__ float sm[];
    float *sk = sm, *sw = sm + nnew * kw;
    const int b = blockIdx.y, j0 = blockIdx.x * 256;
    const float *xb = x + (long)b * N;
    const int base = (j0 / nnew) * norig - width;            // first input sample any output of this block touches
    for (int i = threadIdx.x; i < nnew * kw; i += 256) sk[i] = kern[i];
    for (int i = threadIdx.x; i < nwin; i += 256) {
        const int g = base + i;
        sw[i] = (g >= 0 && g < N) ? xb[g] : 0.f;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= Nout) return;
    const int wdw = j / nnew, ph = j - wdw * nnew;
    const float *kr = sk + ph * kw, *xw = sw + (wdw * norig - width - base);
    float acc = 0.f;
    for (int k = 0; k < kw; ++k) acc = fmaf(kr[k], xw[k], acc);
    out[(long)b * Nout + j] = acc;
}
hipError_t launch_resample_polyphase(const float *x, int B, int N, const float *kern, int norig, int nnew, int width, int kw,
                                     float *out, int Nout, hipStream_t s) {
    const int nwin = (255 / nnew + 1) * norig + kw;          // input window of 256 consecutive outputs
    const size_t lds = ((size_t)nnew * kw + nwin) * sizeof(float);
    if (lds <= 48 * 1024) {   // small rate ratios (16 k -> 22 k: 11 x 22 taps); big ones (44.1 k -> 22 k: 220 x 467) keep the plain kernel
        hipLaunchKernelGGL(resample_polyphase_lds_kernel, dim3((Nout + 255) / 256, B), dim3(256), lds, s, x, N, kern, norig, nnew, width,
                           kw, out, Nout, nwin);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(resample_polyphase_kernel, dim3((Nout + 255) / 256, B), dim3(256), 0, s, x, N, kern, norig, nnew, width,
                       kw, out, Nout);
    return hipGetLastError();
}

// Band-limited interpolation with a table-driven Kaiser-windowed sinc (J. O. Smith's algorithm as published in resampy,
// the resampler behind librosa.load(sr=...) in the reference's pinned librosa 0.9.2: data_utils/utils.py:194).
// win[0..nwin) is the right half of the filter sampled `num_table` times per zero crossing, delta[i] = win[i+1]-win[i].
// Output sample t sits at input time t / ratio; left wing walks x[n], x[n-1], ..., right wing x[n+1], x[n+2], ...
// One thread per output sample; the two table taps of a weight are adjacent floats.
__global__ void resample_kaiser_kernel(const float *__restrict__ x, int N, const float *__restrict__ win,
                                       const float *__restrict__ delta, int nwin, int num_table, double ratio,
                                       float *__restrict__ out, int Nout, int ldo) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Nout) return;
    const float *xb = x + (long)b * N;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(scale * num_table);
    const double time_register = (double)t / ratio;
    const int n = (int)time_register;
    double frac = scale * (time_register - n);
    double index_frac = frac * num_table;
    int offset = (int)index_frac;
    double eta = index_frac - offset;
    int i_max = (nwin - offset) / index_step;
    i_max = i_max < n + 1 ? i_max : n + 1;
    double acc = 0.0;
    for (int i = 0; i < i_max; ++i) {
        const int k = offset + i * index_step;
        acc += ((double)win[k] + eta * (double)delta[k]) * (double)xb[n - i];
    }
    frac = scale - frac;
    index_frac = frac * num_table;
    offset = (int)index_frac;
    eta = index_frac - offset;
    int k_max = (nwin - offset) / index_step;
    k_max = k_max < N - n - 1 ? k_max : N - n - 1;
    for (int k2 = 0; k2 < k_max; ++k2) {
        const int k = offset + k2 * index_step;
        acc += ((double)win[k] + eta * (double)delta[k]) * (double)xb[n + k2 + 1];
    }
    out[(long)b * ldo + t] = (float)(acc * scale);   // resampy scales the filter by the ratio when decimating (unit DC gain)
}
hipError_t launch_resample_kaiser(const float *x, int B, int N, const float *win, const float *delta, int nwin, int num_table,
                                  double ratio, float *out, int Nout, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(resample_kaiser_kernel, dim3((Nout + 255) / 256, B), dim3(256), 0, s, x, N, win, delta, nwin, num_table,
                       ratio, out, Nout, ldo);
    return hipGetLastError();
}

// STFT power spectrum of one frame per workgroup: reflect padding (center=True) + framing + periodic Hann window + a 2048-point real
// FFT + |X|^2, fused (rounds 1-3 ran the transform as a 2048 x 2050 DFT matrix on conv_gemm_f32: 1.26 GMAC per 10 s clip, 150 x the
// arithmetic of an FFT and the largest item of the front-end).  The real transform is a 1024-point complex FFT of z[n] = x[2n] + i x[2n+1]
// — five radix-4 Stockham passes (auto-sorting, no bit reversal) through two LDS buffers, one butterfly per thread per pass, twiddles from
// a table computed in double — followed by the even / odd split X[k] = E[k] - i w^k O[k].  Rounding error grows with log2 N instead of
// sqrt N: closer to the float64 twin than the DFT matrix was.  pw[(b T + t)][0 .. ldp): bins 0 .. 1024, then zeros.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 cmul(f32x2 a, f32x2 b) { return f32x2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]}; }
__global__ __launch_bounds__(256) void stft_power_kernel(const float *__restrict__ x, int N, int T, int hop, const float *__restrict__ win,
                                                         const f32x2 *__restrict__ tw1024, const f32x2 *__restrict__ tw2048,
                                                         float *__restrict__ pw, int ldp) {
    __shared__ f32x2 bufA[1024], bufB[1024];
    const long row = blockIdx.x;   // b * T + t
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const float *xb = x + (long)b * N;
    const int j = threadIdx.x;
    auto sample = [&](int n) {     // windowed sample n of the frame (frame_window of rounds 1-3)
        int i = t * hop + n - 1024;
        if (i < 0) i = -i;
        if (i >= N) i = 2 * (N - 1) - i;
        i = i < 0 ? 0 : (i >= N ? N - 1 : i);
        return xb[i] * win[n];
    };
    auto butterfly = [](f32x2 &v0, f32x2 &v1, f32x2 &v2, f32x2 &v3) {   // 4-point DFT, outputs in natural order
        const f32x2 a = v0 + v2, bb = v0 - v2, c = v1 + v3, d0 = v1 - v3;
        const f32x2 d = {d0[1], -d0[0]};                                  // (v1 - v3) * (-i)
        v0 = a + c; v2 = a - c; v1 = bb + d; v3 = bb - d;
    };
    f32x2 v[4];
    // pass 0 (Ns = 1): inputs straight from the waveform, no twiddles
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = j + 256 * r;
        v[r] = f32x2{sample(2 * n), sample(2 * n + 1)};
    }
    butterfly(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) bufA[j * 4 + r] = v[r];
    __syncthreads();
    // passes 1..4: Ns = 4, 16, 64, 256; twiddle of input r = exp(-2 pi i r (j mod Ns) / (4 Ns)) = tw1024[r (j mod Ns) (256 / Ns)]
    f32x2 *src = bufA, *dst = bufB;
#pragma unroll
    for (int p = 1; p < 5; ++p) {
        const int Ns = 1 << (2 * p), k = j & (Ns - 1), step = 256 >> (2 * p);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = src[j + 256 * r];
            if (r) v[r] = cmul(v[r], tw1024[r * k * step]);
        }
        butterfly(v[0], v[1], v[2], v[3]);
        const int base = ((j - k) << 2) + k;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[base + r * Ns] = v[r];
        __syncthreads();
        f32x2 *tmp = src; src = dst; dst = tmp;
    }
    // src = Z in natural order.  X[k] = E - i w^k O, E = (Z[k] + conj Z[1024 - k]) / 2, O = (Z[k] - conj Z[1024 - k]) / 2, w = exp(-2 pi i / 2048)
    float *out = pw + row * ldp;
    for (int k = j; k < ldp; k += 256) {
        float val = 0.f;
        if (k <= 1024) {
            const f32x2 zk = src[k & 1023], zn0 = src[(1024 - k) & 1023];
            const f32x2 zn = {zn0[0], -zn0[1]};
            const f32x2 e = (zk + zn) * 0.5f, o = (zk - zn) * 0.5f;
            const f32x2 tt = cmul(tw2048[k], o);
            const float xr = e[0] + tt[1], xi = e[1] - tt[0];
            val = xr * xr + xi * xi;
        }
        out[k] = val;
    }
}
hipError_t launch_stft_power(const float *x, int B, int N, int T, int hop, const float *win, const float *tw1024, const float *tw2048,
                             float *pw, int ldp, hipStream_t s) {
    hipLaunchKernelGGL(stft_power_kernel, dim3((unsigned)((long)B * T)), dim3(256), 0, s, x, N, T, hop, win,
                       reinterpret_cast<const f32x2 *>(tw1024), reinterpret_cast<const f32x2 *>(tw2048), pw, ldp);
    return hipGetLastError();
}

// one workgroup per clip: x_db = 10 log10(max(x, 1e-10)); clamp at (clip max - top_db); in place
__global__ __launch_bounds__(256) void db_topdb_kernel(float *__restrict__ mel, long per_clip, float top_db) {
    __shared__ float sm[4];
    float *p = mel + (long)blockIdx.x * per_clip;
    float mx = -INFINITY;
    for (long i = threadIdx.x; i < per_clip; i += 256) {
        const float v = 10.0f * log10f(fmaxf(p[i], 1e-10f));
        p[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const float lo = mx - top_db;
    for (long i = threadIdx.x; i < per_clip; i += 256) p[i] = fmaxf(p[i], lo);
}
hipError_t launch_db_topdb(float *mel, int B, long per_clip, float top_db, hipStream_t s) {
    hipLaunchKernelGGL(db_topdb_kernel, dim3(B), dim3(256), 0, s, mel, per_clip, top_db);
    return hipGetLastError();
}

}  // namespace ts
