// Device audio front-end kernels (everything GEMM-shaped — the DFT, the mel projection, the DCT — runs on conv_gemm_f32):
//   resample_polyphase   torchaudio.transforms.Resample (sinc_interp_hann) as a polyphase FIR
//   frame_window         reflect padding (center=True) + framing (n_fft, hop) + periodic Hann window
//   power_spectrum       |X|^2 from the interleaved (re, im) DFT output, zero-padded to a multiple of 32 bins
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
hipError_t launch_resample_polyphase(const float *x, int B, int N, const float *kern, int norig, int nnew, int width, int kw,
                                     float *out, int Nout, hipStream_t s) {
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

// frames[(b*T + t)][n] = w[n] * x_b[reflect(t*hop + n - n_fft/2)]
__global__ void frame_window_kernel(const float *__restrict__ x, int N, int T, int hop, int nfft, const float *__restrict__ win,
                                    float *__restrict__ frames) {
    const long row = blockIdx.x;   // b*T + t
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const float *xb = x + (long)b * N;
    for (int n = threadIdx.x; n < nfft; n += blockDim.x) {
        int i = t * hop + n - nfft / 2;
        if (i < 0) i = -i;
        if (i >= N) i = 2 * (N - 1) - i;
        i = i < 0 ? 0 : (i >= N ? N - 1 : i);
        frames[row * nfft + n] = xb[i] * win[n];
    }
}
hipError_t launch_frame_window(const float *x, int B, int N, int T, int hop, int nfft, const float *win, float *frames,
                               hipStream_t s) {
    hipLaunchKernelGGL(frame_window_kernel, dim3((unsigned)((long)B * T)), dim3(256), 0, s, x, N, T, hop, nfft, win, frames);
    return hipGetLastError();
}

__global__ void power_spectrum_kernel(const float *__restrict__ spec, int lds_, int nbins, float *__restrict__ pw, int ldp,
                                      long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldp) return;
    const long m = i / ldp;
    const int f = (int)(i - m * ldp);
    float v = 0.f;
    if (f < nbins) {
        const float re = spec[m * lds_ + 2 * f], im = spec[m * lds_ + 2 * f + 1];
        v = re * re + im * im;
    }
    pw[i] = v;
}
hipError_t launch_power_spectrum(const float *spec, int lds_, int nbins, float *pw, int ldp, long rows, hipStream_t s) {
    const long n = rows * ldp;
    hipLaunchKernelGGL(power_spectrum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, spec, lds_, nbins, pw, ldp, rows);
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
